#!/usr/bin/env python
"""Condense rocprofv3 output directories into the small files kept under profiles/.

    python profiles/summarise.py stats  <rocprof dir> <out.csv>            # kernel_stats.csv of a --kernel-trace --stats run
    python profiles/summarise.py pmc    <out.json> <rocprof dir> [...]     # counter_collection.csv of --pmc runs
    python profiles/summarise.py pmc    <out.json> --grid 65536 <rocprof dir> [...]   # only launches of that many threads
(round 3: bench.py builds its frontier with real branch-and-bound searches -- hundreds of small launches of the 2- and
4-wave kernels before the timed ones; --grid keeps the launches of the timed kernel: 1024 workgroups x 64 threads)

For counters the mean per launch of `hmpc_qp_kernel` (and, under 'shift_kernel', of `hmpc_shift_kernel`) is stored; launches that belong to the
warm-up are included (the kernel does the same work in each).  FETCH_SIZE / WRITE_SIZE are in
KiB as rocprofv3 reports them; `*_bytes_corrected` applies the gfx950 correction of
MI355X_MICROARCH.md (section HBM): FETCH_SIZE tallies 128-B requests at 64 B => x2; WRITE_SIZE exact.
"""
import csv
import glob
import json
import os
import shutil
import sys


def find(d, suffix):
    """Files of the NEWEST run in the directory: gpurun merges the box's output into the local directory of the same
    name, so a directory that was collected twice holds both runs (one process-id prefix each)."""
    hits = sorted(glob.glob(os.path.join(d, '**', '*' + suffix), recursive=True), key=os.path.getmtime)
    if not hits:
        raise SystemExit('no *%s under %s' % (suffix, d))
    return hits[-1:]


def stats(d, out):
    shutil.copyfile(find(d, '_kernel_stats.csv')[0], out)


GRID = None


def pmc(out, dirs):
    res = _pmc(dirs, 'hmpc_qp_kernel', grid=GRID)
    shift = _pmc(dirs, 'hmpc_shift_kernel', required=False)
    if shift:
        res['shift_kernel'] = shift      # the warm-start shift (bench.py: 65536 leaves per launch)
    with open(out, 'w') as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


def _pmc(dirs, kernel, required=True, grid=None):
    acc = {}
    for d in dirs:
        for f in find(d, '_counter_collection.csv'):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if kernel not in row['Kernel_Name']:
                        continue
                    if grid is not None and int(row['Grid_Size']) != grid:
                        continue
                    a = acc.setdefault(row['Counter_Name'], {'sum': 0.0, 'launches': 0, 'kernel': row['Kernel_Name'],
                                                             'vgpr': row['VGPR_Count'], 'agpr': row['Accum_VGPR_Count'],
                                                             'sgpr': row['SGPR_Count'], 'lds': row['LDS_Block_Size'],
                                                             'scratch': row['Scratch_Size'], 'grid': row['Grid_Size'],
                                                             'wg': row['Workgroup_Size']})
                    a['sum'] += float(row['Counter_Value'])
                    a['launches'] += 1
    res = {}
    for k, a in sorted(acc.items()):
        res[k] = {'mean_per_launch': a['sum'] / a['launches'], 'launches': a['launches']}
        res.setdefault('_kernel', {kk: a[kk] for kk in ('kernel', 'vgpr', 'agpr', 'sgpr', 'lds', 'scratch', 'grid', 'wg')})
    if 'FETCH_SIZE' in res:
        res['FETCH_SIZE']['unit'] = 'KiB'
        res['FETCH_SIZE']['bytes_corrected'] = res['FETCH_SIZE']['mean_per_launch'] * 1024 * 2
    if 'WRITE_SIZE' in res:
        res['WRITE_SIZE']['unit'] = 'KiB'
        res['WRITE_SIZE']['bytes_corrected'] = res['WRITE_SIZE']['mean_per_launch'] * 1024
    if 'FETCH_SIZE' in res and 'WRITE_SIZE' in res:
        res['hbm_traffic_bytes_per_launch'] = res['FETCH_SIZE']['bytes_corrected'] + res['WRITE_SIZE']['bytes_corrected']
    return res


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        stats(sys.argv[2], sys.argv[3])
    else:
        rest = sys.argv[3:]
        if rest and rest[0] == '--grid':
            GRID = int(rest[1])
            rest = rest[2:]
        pmc(sys.argv[2], rest)
