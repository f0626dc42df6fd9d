"""Lists the short dispatches (< 30 us) on the main queue of the last train step in a rocprofv3
kernel trace of tools/fixed_step.py: each costs ~6 us of serialised launch latency."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
ends = [i for i, r in enumerate(rows) if 'adadelta' in r['Kernel_Name']]
step = rows[ends[-2] + 1:ends[-1] + 1]
t0 = rows[ends[-2]]['e']
main = rows[ends[-1]]['Queue_Id']
tiny = n = 0
for r in step:
    d = (r['e'] - r['s']) / 1e3
    nm = re.sub(r'\(anonymous namespace\)::|at::native::', '', r['Kernel_Name'])
    if d < 30 and r['Queue_Id'] == main:
        tiny += d; n += 1
        print('%8.1f %5.1f  grid %-8s %s' % ((r['s'] - t0) / 1e3, d, r['Grid_Size_X'], nm[:130]))
print('step span %.3f ms; %d short dispatches on the main queue, %.0f us busy' % ((step[-1]['e'] - t0) / 1e6, n, tiny))
