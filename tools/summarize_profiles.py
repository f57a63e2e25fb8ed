"""Condense rocprofv3 output directories into the small files kept under profiles/.

usage: summarize_profiles.py <tag> <trace_dir> <fetch_dir> <write_dir> [bench_json]
  trace_dir: rocprofv3 --kernel-trace --stats -d <trace_dir> -- python3 bench.py ...
  fetch_dir: rocprofv3 --pmc FETCH_SIZE -d <fetch_dir> -- python3 bench.py ...   (own pass)
  write_dir: rocprofv3 --pmc WRITE_SIZE -d <write_dir> -- python3 bench.py ...   (own pass)
Writes profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc_traffic.json and copies the bench line
printed under the profiler to profiles/<tag>_bench_under_rocprof.json.
"""
import csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, '**', '*' + suffix), recursive=True))
    return hits[0] if hits else None


def counter_avg(d, name):
    path = find(d, '_counter_collection.csv')
    out = {}
    if not path:
        return out
    with open(path) as f:
        for row in csv.DictReader(f):
            if row['Counter_Name'] != name:
                continue
            k = out.setdefault(row['Kernel_Name'], [0, 0.0])
            k[0] += 1
            k[1] += float(row['Counter_Value'])
    return {k: {'launches': v[0], 'avg': v[1] / v[0]} for k, v in out.items()}


def main():
    tag, trace_dir, fetch_dir, write_dir = sys.argv[1:5]
    prof = os.path.join(ROOT, 'profiles')
    os.makedirs(prof, exist_ok=True)
    stats = find(trace_dir, '_kernel_stats.csv')
    if stats:
        shutil.copy(stats, os.path.join(prof, tag + '_kernel_stats.csv'))
    fetch, write = counter_avg(fetch_dir, 'FETCH_SIZE'), counter_avg(write_dir, 'WRITE_SIZE')
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        e = {}
        if k in fetch:
            e['FETCH_SIZE_KB'] = fetch[k]
        if k in write:
            e['WRITE_SIZE_KB'] = write[k]
        # gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> x2; WRITE_SIZE exact
        e['hbm_bytes_per_launch_corrected'] = (2.0 * fetch.get(k, {'avg': 0})['avg'] + write.get(k, {'avg': 0})['avg']) * 1024.0
        kernels[k] = e
    note = ('rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over `python3 bench.py --steps 3 '
            '--warmup 1 --no-cpu-baseline` on MI355X; Counter_Value is in KB; on gfx950 FETCH_SIZE counts 64 B per '
            '128-B request for wide coalesced reads, so read bytes = 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM '
            'section); WRITE_SIZE is exact')
    with open(os.path.join(prof, tag + '_pmc_traffic.json'), 'w') as f:
        json.dump({'note': note, 'kernels': kernels}, f, indent=1)
    if len(sys.argv) > 5 and os.path.exists(sys.argv[5]):
        with open(sys.argv[5]) as f:
            lines = [l for l in f if l.startswith('{')]
        if lines:
            with open(os.path.join(prof, tag + '_bench_under_rocprof.json'), 'w') as f:
                f.write(lines[-1])


if __name__ == '__main__':
    main()
