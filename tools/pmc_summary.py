"""Turn two `rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace --output-format csv` passes into
profiles/rNN_pmc_traffic_<workload>.json: HBM bytes per launch and kernel, corrected as MI355X_MICROARCH.md's
HBM/rocprofv3 section prescribes for gfx950 (FETCH_SIZE counts 32-B units reported in KiB at half weight ->
doubled; WRITE_SIZE as is).  Usage: pmc_summary.py <dir_FETCH> <dir_WRITE> <out.json>"""
import csv, glob, hashlib, json, os, re, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sources_sha():
    """sha256 over the kernel sources: bench.py marks `traffic` stale when the library was rebuilt from other sources."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'end-to-end-asr-pytorch_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h')):
            h.update(f.encode() + b'\0' + open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


def load(d, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') != counter:
                continue
            k = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')
            k = re.sub(r'\(.*$', '', k)
            acc[k][0] += float(r['Counter_Value'])
            acc[k][1] += 1
    return acc


def main():
    fd, wd, out = sys.argv[1:4]
    fe, wr = load(fd, 'FETCH_SIZE'), load(wd, 'WRITE_SIZE')
    res = {}
    for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, [0, 1])[0] + wr.get(k, [0, 1])[0])):
        n = max(fe.get(k, [0, 0])[1], wr.get(k, [0, 0])[1])
        if n == 0:
            continue
        f_kib = fe.get(k, [0, 1])[0] / max(1, fe.get(k, [0, 1])[1])
        w_kib = wr.get(k, [0, 1])[0] / max(1, wr.get(k, [0, 1])[1])
        res[k] = dict(launches=n, fetch_KiB_per_launch=f_kib, write_KiB_per_launch=w_kib,
                      hbm_MB_per_launch_corrected=(2 * f_kib + w_kib) * 1024 / 1e6)
    top = list(res.items())[:12]
    res['_meta'] = dict(kernel_sources_sha=kernel_sources_sha())
    json.dump(res, open(out, 'w'), indent=1)
    for k, v in top:
        print(f"{k[:60]:60s} n={v['launches']:5d}  {v['hbm_MB_per_launch_corrected']:10.2f} MB/launch")


if __name__ == '__main__':
    main()
