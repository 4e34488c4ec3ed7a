"""Timeline of the per-substep launch mode from a rocprofv3 kernel trace:
start / end of consecutive substep_multi_kernel dispatches per queue, overlap
between the two half-ensemble chains.
  cd /tmp && rocprofv3 --kernel-trace -d DIR -o run -- python bench.py --launch-mode per_substep ...
  python profiles/tools/substep_overlap_trace.py DIR
"""
import csv, glob, sys
import numpy as np
path = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(path)) if 'substep_multi' in r['Kernel_Name']]
start = np.array([int(r['Start_Timestamp']) for r in rows], dtype=np.int64)
end = np.array([int(r['End_Timestamp']) for r in rows], dtype=np.int64)
queue = np.array([int(r['Queue_Id']) for r in rows])
order = np.argsort(start)
start, end, queue = start[order], end[order], queue[order]
n = len(start)
mid = slice(n // 2, n // 2 + 16)
print('dispatches', n, 'queues', sorted(set(queue.tolist())))
print('mean duration per dispatch (us)', (end - start).mean() / 1e3)
t0 = start[mid][0]
for s, e, q in zip(start[mid], end[mid], queue[mid]):
  print('  queue %d  start %8.2f us  end %8.2f us  dur %6.2f' % (q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
# busy fraction: union of intervals over the steady-state window
lo, hi = n // 4, 3 * n // 4
ev = sorted([(s, 1) for s in start[lo:hi]] + [(e, -1) for e in end[lo:hi]])
depth, last, t_two, t_one, t_zero = 0, ev[0][0], 0, 0, 0
for t, d in ev:
  if depth >= 2: t_two += t - last
  elif depth == 1: t_one += t - last
  else: t_zero += t - last
  depth += d; last = t
tot = t_two + t_one + t_zero
print('time with 2 / 1 / 0 kernels in flight: %.1f %% / %.1f %% / %.1f %%' % (100 * t_two / tot, 100 * t_one / tot, 100 * t_zero / tot))
print('wall per substep pair (us):', (end[hi - 1] - start[lo]) / 1e3 / ((hi - lo) / 2))
