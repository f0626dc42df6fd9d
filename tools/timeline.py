"""Reads a rocprofv3 --kernel-trace CSV of tools/fixed_step.py and prints the timeline of the
last complete train step: per kernel (merged runs of the same kernel on the same queue) the
start offset, busy time, count and the idle gap in front of it on its queue."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
ends = [i for i, r in enumerate(rows) if 'adadelta' in r['Kernel_Name']]
a, b = ends[-2] + 1, ends[-1] + 1
step = rows[a:b]
t0 = rows[ends[-2]]['e']
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*', '', n)
    n = re.sub(r'at::native::', '', n)
    return n[:60]
print('step span %.3f ms, %d dispatches' % ((step[-1]['e'] - t0) / 1e6, len(step)))
queues = sorted({r['Queue_Id'] for r in step})
busy = {q: 0 for q in queues}
last_end = {q: t0 for q in queues}
out = []
cur = None
for r in step:
    q = r['Queue_Id']; n = short(r['Kernel_Name'])
    gap = r['s'] - last_end[q]
    busy[q] += r['e'] - r['s']
    if cur and cur['q'] == q and cur['n'] == n and gap < 20000:
        cur['busy'] += r['e'] - r['s']; cur['cnt'] += 1; cur['end'] = r['e']; cur['gaps'] += max(gap, 0)
    else:
        cur = {'q': q, 'n': n, 'start': r['s'], 'busy': r['e'] - r['s'], 'cnt': 1, 'end': r['e'], 'gap': gap, 'gaps': 0}
        out.append(cur)
    last_end[q] = max(last_end[q], r['e'])
for c in out:
    print('%9.1f us  q%-2s %-60s x%-3d busy %8.1f us  span %8.1f  gap before %7.1f' %
          ((c['start'] - t0) / 1e3, c['q'], c['n'], c['cnt'], c['busy'] / 1e3, (c['end'] - c['start']) / 1e3, c['gap'] / 1e3))
for q in queues:
    print('queue %s busy %.3f ms' % (q, busy[q] / 1e6))
