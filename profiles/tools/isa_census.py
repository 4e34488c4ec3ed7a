#!/usr/bin/env python3
"""Static census of the gfx950 code objects in csrc/build*/ (no GPU needed).

  python profiles/tools/isa_census.py notes  [--dir csrc/build] [--match substr]
      per kernel: VGPRs, AGPRs, SGPRs, VGPR / SGPR spills, scratch bytes, LDS bytes
      (the AMDGPU metadata note of every object, llvm-readelf --notes)
  python profiles/tools/isa_census.py census OBJECT KERNEL_SUBSTR [--dump file]
      instruction census of the kernel's hottest loop (the innermost backward branch
      span that holds the most MFMAs): instruction classes, and the VALU
      instructions by mnemonic.  --dump writes the loop with MFMA runs collapsed.

The numbers in profiles/r5_valu_census.txt and profiles/r5_spill_table.txt come
from this script.
"""
import argparse
import collections
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, 'data-driven-discretization-1d_amd', 'csrc')
LLVM = '/opt/rocm/lib/llvm/bin'


def extract(obj, workdir):
  """The gfx950 code object bundled in a host object / shared library."""
  local = os.path.join(workdir, os.path.basename(obj))
  shutil.copy(obj, local)
  subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--offloading', local],
                 check=True, capture_output=True, cwd=workdir)
  outs = sorted(glob.glob(local + '.*gfx950'))
  if not outs:
    raise RuntimeError('no gfx950 code object in ' + obj)
  return outs


def demangle(names):
  done = subprocess.run(['c++filt'], input='\n'.join(names),
                        capture_output=True, text=True)
  if done.returncode != 0:
    return names
  return done.stdout.split('\n')[:len(names)]


def short(name):
  name = re.sub(r'^void ', '', name)
  name = re.sub(r'\(ddd::.*$', '', name)
  name = name.replace('ddd::', '').replace('(bool)', '').replace('(int)', '')
  return name


def kernel_notes(code_object):
  text = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', code_object],
                        check=True, capture_output=True, text=True).stdout
  kernels = []
  cur = None
  for line in text.split('\n'):
    m = re.match(r'\s*-?\s*\.(\w+):\s*(.*)$', line)
    if not m:
      continue
    key, val = m.group(1), m.group(2).strip()
    if key == 'agpr_count' or (key == 'args' and cur is None):
      pass
    if key == 'name' and val.startswith('_Z') and not val.endswith('.kd'):
      # (.name appears for arguments too: kernel names are mangled symbols)
      if cur is not None and 'symbol' not in cur:
        cur['name'] = val
      continue
    if key in ('agpr_count', 'group_segment_fixed_size', 'private_segment_fixed_size',
               'sgpr_count', 'sgpr_spill_count', 'vgpr_count', 'vgpr_spill_count',
               'max_flat_workgroup_size', 'symbol', 'kernarg_segment_size'):
      if key == 'agpr_count':
        cur = {}
        kernels.append(cur)
      if cur is not None:
        cur[key] = val
  return [k for k in kernels if 'symbol' in k]


def cmd_notes(args):
  objs = sorted(glob.glob(os.path.join(args.dir, '*.o')))
  rows = []
  with tempfile.TemporaryDirectory() as work:
    for obj in objs:
      for co in extract(obj, work):
        for k in kernel_notes(co):
          rows.append((os.path.basename(obj), k))
  names = demangle([k['symbol'].replace('.kd', '') for _, k in rows])
  print('{:<28s} {:>5s} {:>5s} {:>5s} {:>6s} {:>6s} {:>8s} {:>7s}  kernel'.format(
      'object', 'VGPR', 'AGPR', 'SGPR', 'vspill', 'sspill', 'scratch', 'LDS'))
  for (obj, k), name in zip(rows, names):
    if args.match and args.match not in name:
      continue
    print('{:<28s} {:>5s} {:>5s} {:>5s} {:>6s} {:>6s} {:>8s} {:>7s}  {}'.format(
        obj, k.get('vgpr_count', '?'), k.get('agpr_count', '?'), k.get('sgpr_count', '?'),
        k.get('vgpr_spill_count', '?'), k.get('sgpr_spill_count', '?'),
        k.get('private_segment_fixed_size', '?'), k.get('group_segment_fixed_size', '?'),
        short(name)))


def classify(op):
  if op.startswith('v_mfma'):
    return 'mfma'
  if op.startswith('v_'):
    return 'valu'
  if op.startswith('ds_'):
    return 'lds'
  if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
    return 'vmem'
  if op.startswith('s_load') or op.startswith('s_buffer_load'):
    return 'smem'
  if op.startswith('s_waitcnt'):
    return 'waitcnt'
  if op.startswith('s_nop'):
    return 'nop'
  if op.startswith(('s_cbranch', 's_branch')):
    return 'branch'
  if op.startswith('s_barrier'):
    return 'barrier'
  return 'salu'


def disassemble(code_object):
  text = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--no-show-raw-insn',
                         code_object], check=True, capture_output=True, text=True).stdout
  kernels = collections.OrderedDict()
  cur = None
  for line in text.split('\n'):
    m = re.match(r'^[0-9a-f]+ <(\S+)>:', line)
    if m:
      cur = []
      kernels[m.group(1)] = cur
      continue
    m = re.match(r'^\s+(\S+)\s*(.*?)\s*// ([0-9A-F]+):', line)
    if m and cur is not None:
      cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
  return kernels


def hottest_loop(insns):
  """(first, last) indices of the innermost backward-branch span with the most MFMAs."""
  addr_index = {a: i for i, (a, _, _) in enumerate(insns)}
  best = None
  for i, (addr, op, operands) in enumerate(insns):
    if not op.startswith(('s_cbranch', 's_branch')):
      continue
    off = int(operands.split()[0])
    if off >= 32768:
      off -= 65536
    target = addr + 4 + 4 * off
    if target > addr or target not in addr_index:
      continue
    j = addr_index[target]
    mf = sum(1 for _, o, _ in insns[j:i + 1] if o.startswith('v_mfma'))
    if mf == 0:
      continue
    # innermost = the shortest span among those holding the (near-)maximal MFMA count
    key = (-mf, i - j)
    if best is None or key < best[0]:
      best = (key, j, i)
  if best is None:
    return 0, len(insns) - 1
  # prefer the SHORTEST loop that still holds >= 90 % of the best MFMA count
  top = -best[0][0]
  cands = []
  for i, (addr, op, operands) in enumerate(insns):
    if not op.startswith(('s_cbranch', 's_branch')):
      continue
    off = int(operands.split()[0])
    if off >= 32768:
      off -= 65536
    target = addr + 4 + 4 * off
    if target > addr or target not in addr_index:
      continue
    j = addr_index[target]
    mf = sum(1 for _, o, _ in insns[j:i + 1] if o.startswith('v_mfma'))
    if mf >= 0.9 * top:
      cands.append((i - j, j, i))
  cands.sort()
  return cands[0][1], cands[0][2]


def cmd_census(args):
  with tempfile.TemporaryDirectory() as work:
    kernels = collections.OrderedDict()
    for co in extract(args.object, work):
      kernels.update(disassemble(co))
  names = list(kernels)
  pretty = demangle(names)
  hits = [(n, p) for n, p in zip(names, pretty) if args.kernel in p]
  if not hits:
    sys.exit('no kernel matches {!r}; have:\n  '.format(args.kernel) + '\n  '.join(pretty))
  for name, p in hits:
    insns = kernels[name]
    lo, hi = hottest_loop(insns)
    loop = insns[lo:hi + 1]
    classes = collections.Counter(classify(op) for _, op, _ in loop)
    valu = collections.Counter(re.sub(r'_e32$|_e64$|_dpp$|_sdwa$', '', op)
                               for _, op, _ in loop if classify(op) == 'valu')
    print('kernel:', short(p))
    print('  whole kernel: {} instructions; hottest loop: {} instructions ({:#x}..{:#x})'.format(
        len(insns), len(loop), loop[0][0], loop[-1][0]))
    print('  classes:', ', '.join('{} {}'.format(k, v) for k, v in sorted(classes.items())))
    print('  VALU by mnemonic ({} total):'.format(sum(valu.values())))
    for op, cnt in valu.most_common():
      print('    {:5d}  {}'.format(cnt, op))
    if args.dump:
      with open(args.dump, 'w') as f:
        run = 0
        for _, op, operands in loop:
          if op.startswith('v_mfma'):
            run += 1
            continue
          if run:
            f.write('    ... {} mfma\n'.format(run))
            run = 0
          f.write('{} {}\n'.format(op, operands))
        if run:
          f.write('    ... {} mfma\n'.format(run))


def main():
  ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
  sub = ap.add_subparsers(dest='cmd', required=True)
  n = sub.add_parser('notes')
  n.add_argument('--dir', default=os.path.join(CSRC, 'build'))
  n.add_argument('--match', default='')
  n.set_defaults(fn=cmd_notes)
  c = sub.add_parser('census')
  c.add_argument('object')
  c.add_argument('kernel')
  c.add_argument('--dump', default='')
  c.set_defaults(fn=cmd_census)
  args = ap.parse_args()
  args.fn(args)


if __name__ == '__main__':
  main()
