"""Copies a rocprofv3 --kernel-trace --stats output into profiles/ as a small
CSV plus a markdown table.  usage: summarize_profile.py <rocprof dir> <tag> <steps>"""
import csv
import glob
import os
import re
import sys

src, tag, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stats = max(glob.glob(os.path.join(src, '*', '*_kernel_stats.csv')), key=os.path.getmtime)     # the newest run in the directory
rows = list(csv.DictReader(open(stats)))
out_csv = os.path.join(root, 'profiles', tag + '_kernel_stats.csv')
with open(out_csv, 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
    for r in rows:
        w.writerow([r['Name'], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'],
                    r['MinNs'], r['MaxNs']])
tot = sum(float(r['TotalDurationNs']) for r in rows)
with open(os.path.join(root, 'profiles', tag + '_summary.md'), 'w') as f:
    f.write('# %s: rocprofv3 --kernel-trace --stats, %d train steps profiled\n\n' % (tag, steps))
    f.write('Total kernel time %.2f ms = %.2f ms per step.\n\n' % (tot / 1e6, tot / 1e6 / steps))
    f.write('| kernel | calls | avg us | min us | total ms | % |\n|---|---|---|---|---|---|\n')
    for r in rows[:24]:
        name = re.sub(r'\(anonymous namespace\)::', '', r['Name'])
        name = re.sub(r'\(.*', '', name)[:60]
        f.write('| `%s` | %s | %.2f | %.2f | %.2f | %.1f |\n' % (
            name, r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3,
            float(r['TotalDurationNs']) / 1e6, 100 * float(r['TotalDurationNs']) / tot))
print(open(os.path.join(root, 'profiles', tag + '_summary.md')).read())
