"""Turn a rocprofv3 rocpd sqlite result (default output of `rocprofv3 --kernel-trace --stats`) into
a small text summary that can be committed under profiles/.

  python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_x_kernel_stats.txt
"""
import sqlite3
import sys


def main(path, top=12):
  c = sqlite3.connect(path)
  rows = c.execute('select name, total_calls, total_duration, average, percentage from top_kernels').fetchall()
  print(f'# rocprofv3 --kernel-trace --stats summary of {path}')
  print('# durations in microseconds')
  print(f'{"calls":>7} {"total_us":>12} {"avg_us":>10} {"pct":>7}  kernel')
  for name, calls, total, avg, pct in rows[:top]:
    short = name if len(name) < 110 else name[:107] + '...'
    print(f'{calls:7d} {total:12.2f} {avg:10.3f} {pct:7.2f}  {short}')
  try:
    rows = c.execute('select name, min(duration), max(duration), avg(duration), count(*), max(vgpr_count), '
                     'max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels '
                     "where name like '%step_kernel%' or name like '%small_obs%' or name like '%rollout%' "
                     'group by name').fetchall()
    print('\n# engine kernels: min/max/avg duration (ns), launches, vgpr, sgpr, lds bytes, grid_x, wg_x')
    for r in rows:
      print(' ', r)
  except sqlite3.Error as e:  # pragma: no cover
    print('# (kernel detail unavailable:', e, ')')


if __name__ == '__main__':
  main(sys.argv[1])
