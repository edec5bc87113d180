#!/usr/bin/env python3
"""Timeline of ONE train step from a `rocprofv3 --kernel-trace` csv (tools/r2_trace.sh): every dispatch with its start
relative to the step, duration, queue, and the idle gap on the busiest queue; plus per-queue busy time and the time
during which only side-queue work runs.  A step = the dispatches between two `adadelta_kernel` / `adam_kernel` ends.

  python tools/timeline.py gpurun_out/trace_x/**/t_kernel_trace.csv [--step 5] [--min-us 15]
"""
import argparse
import csv
import re
import sys


def short(name):
    m = re.search(r'(\w+)<([^>]*)>', name)
    if m:
        return f'{m.group(1)}<{m.group(2)[:28]}>'
    m = re.search(r'(\w+)\(', name)
    return (m.group(1) if m else name)[:50]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('csv')
    ap.add_argument('--step', type=int, default=5)
    ap.add_argument('--min-us', type=float, default=15.0)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0'), r.get('Stream_Id', '0'))
                 for r in rows), key=lambda e: e[0])
    ends = [e[1] for e in ev if 'adadelta_kernel' in e[2] or 'adam_kernel' in e[2]]
    if len(ends) <= a.step:
        sys.exit(f'only {len(ends)} optimiser launches in the trace')
    t0, t1 = ends[a.step - 1], ends[a.step]
    step = [e for e in ev if e[0] >= t0 and e[1] <= t1 + 1]
    print(f'step {a.step}: {(t1 - t0) / 1e6:.3f} ms, {len(step)} dispatches')
    queues = {}
    for e in step:
        queues.setdefault((e[3], e[4]), []).append(e)
    main_q = max(queues, key=lambda q: sum(e[1] - e[0] for e in queues[q]))
    for q, es in sorted(queues.items(), key=lambda kv: -sum(e[1] - e[0] for e in kv[1])):
        print(f'  queue {q}: {len(es)} dispatches, busy {sum(e[1] - e[0] for e in es) / 1e6:.3f} ms' + ('  <- main' if q == main_q else ''))
    # union of busy intervals
    def union(es):
        tot, cur_s, cur_e = 0, None, None
        for s, e_, *_ in sorted(es):
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    tot += cur_e - cur_s
                cur_s, cur_e = s, e_
            else:
                cur_e = max(cur_e, e_)
        return tot + (cur_e - cur_s if cur_e is not None else 0)
    print(f'  GPU busy (any queue) {union(step) / 1e6:.3f} ms; main queue busy {union(queues[main_q]) / 1e6:.3f} ms')
    print('  --- main queue, dispatches >= %.0f us or gaps >= 10 us (t in ms from step start) ---' % a.min_us)
    prev_end = t0
    small_t, small_n = 0, 0
    for s, e_, name, *_ in sorted(queues[main_q]):
        gap = s - prev_end
        if gap >= 10_000:
            print(f'  {(prev_end - t0) / 1e6:8.3f}   ... idle {gap / 1e3:7.1f} us')
        d = e_ - s
        if d >= a.min_us * 1e3:
            if small_n:
                print(f'             ({small_n} small dispatches, {small_t / 1e3:.1f} us)')
                small_t, small_n = 0, 0
            print(f'  {(s - t0) / 1e6:8.3f}  {d / 1e3:8.1f} us  {short(name)}')
        else:
            small_t += d
            small_n += 1
        prev_end = max(prev_end, e_)
    if small_n:
        print(f'             ({small_n} small dispatches, {small_t / 1e3:.1f} us)')
    for q, es in queues.items():
        if q == main_q:
            continue
        print(f'  --- queue {q} ---')
        for s, e_, name, *_ in sorted(es):
            if e_ - s >= a.min_us * 1e3:
                print(f'  {(s - t0) / 1e6:8.3f}  {(e_ - s) / 1e3:8.1f} us  {short(name)}')


if __name__ == '__main__':
    main()
