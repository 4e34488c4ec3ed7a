#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace --stats) as text.

rocprofv3 on this image writes <name>_results.db instead of CSV by default;
this prints the per-kernel totals (view `top_kernels`) and every dispatch of
the ddd kernels with its launch geometry and register use.

    python profiles/summarize_rocpd.py gpurun_out/prof_r1/bench_results.db
"""
import sqlite3
import sys


def main(path):
  con = sqlite3.connect(path)
  cur = con.cursor()
  print('# source: {}'.format(path))
  print('# per-kernel totals (durations in microseconds)')
  print('{:>6} {:>14} {:>14} {:>7}  {}'.format('calls', 'total_us', 'avg_us',
                                                 'pct', 'kernel'))
  for name, calls, total, avg, pct in cur.execute(
      'select name, total_calls, total_duration, average, percentage '
      'from top_kernels order by total_duration desc'):
    print('{:>6} {:>14.3f} {:>14.3f} {:>7.3f}  {}'.format(
        calls, total, avg, pct, name[:110]))
  print()
  print('# ddd kernel dispatches')
  print('{:>12} {:>8} {:>6} {:>8} {:>6} {:>6} {:>6}  {}'.format(
      'duration_us', 'grid_x', 'wg_x', 'lds', 'vgpr', 'agpr', 'sgpr', 'kernel'))
  for row in cur.execute(
      'select name, duration, grid_x, workgroup_x, lds_size, vgpr_count, '
      'accum_vgpr_count, sgpr_count from kernels where name like "%ddd::%" '
      'order by start'):
    name, dur, gx, wx, lds, vg, ag, sg = row
    print('{:>12.3f} {:>8} {:>6} {:>8} {:>6} {:>6} {:>6}  {}'.format(
        dur / 1e3, gx, wx, lds, vg, ag, sg, name[:90]))


if __name__ == '__main__':
  main(sys.argv[1])
