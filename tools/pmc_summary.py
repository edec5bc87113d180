"""Turn two `rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace --output-format csv` passes into
profiles/rNN_pmc_traffic_<workload>.json: HBM bytes per launch and kernel, corrected as MI355X_MICROARCH.md's
HBM/rocprofv3 section prescribes for gfx950 (FETCH_SIZE counts 32-B units reported in KiB at half weight ->
doubled; WRITE_SIZE as is).  With a fourth argument, the directory of a `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
GRBM_GUI_ACTIVE` pass, every kernel also gets its MFMA utilisation: mfma_busy_cycles per launch (the SIMDs' matrix pipes,
summed over the chip), the launch's wall cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs -> / 8) and
mfma_util = mfma_busy / (wall cycles x 1024 SIMDs), the fraction of the chip's matrix-pipe cycles the launch used.
Usage: pmc_summary.py <dir_FETCH> <dir_WRITE> <out.json> [<dir_MFMA>]"""
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
    if len(sys.argv) > 4:
        mf, bz, gui = (load(sys.argv[4], c) for c in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'GRBM_GUI_ACTIVE'))
        for k in set(mf) | set(gui):
            n = max(mf.get(k, [0, 0])[1], gui.get(k, [0, 0])[1])
            if n == 0:
                continue
            m = mf.get(k, [0, 1])[0] / max(1, mf.get(k, [0, 1])[1])
            w = gui.get(k, [0, 1])[0] / max(1, gui.get(k, [0, 1])[1]) / 8.0
            b = bz.get(k, [0, 1])[0] / max(1, bz.get(k, [0, 1])[1])
            r = res.setdefault(k, dict(launches=n))
            r.update(mfma_busy_cycles_per_launch=m, wall_cycles_per_launch=w, sq_busy_cycles_per_launch=b,
                     mfma_util=(m / (w * 1024.0) if w > 0 else None))
    top = list(res.items())[:12]
    res['_meta'] = dict(kernel_sources_sha=kernel_sources_sha())
    json.dump(res, open(out, 'w'), indent=1)
    for k, v in top:
        mu = v.get('mfma_util')
        print(f"{k[:60]:60s} n={v['launches']:5d}  {v.get('hbm_MB_per_launch_corrected', 0.0):10.2f} MB/launch" +
              (f"  mfma_util {mu:.4f}" if mu is not None else ''))


if __name__ == '__main__':
    main()
