"""Summarise a rocprofv3 kernel trace (CSV) of bench.py: how many kernels run at a time, and what the GPU spends its time on.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/pr -o kt -- python3 bench.py ...   (on the GPU box)
    python3 tools/trace_overlap.py /tmp/pr/kt_kernel_trace.csv out.json
"""
import csv
import json
import sys
from collections import defaultdict


def klass(name):
    for key, lab in (('decoder_self_attention', 'self_attention'), ('decoder_cross_attention', 'cross_attention'),
                     ('gemm_split', 'gemm'), ('gemm_f32', 'gemm'), ('add_layer_norm', 'layer_norm'), ('log_softmax_topk', 'topk'),
                     ('beam_step', 'beam_step'), ('encoder_attention', 'encoder_attention')):
        if key in name:
            return lab
    return 'other'


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), klass(r['Kernel_Name'])))
    rows.sort()
    # keep the busiest contiguous stretch: drop the leading 25 % and trailing 25 % of kernels (warm-up, profile legs)
    lo, hi = len(rows) // 4, len(rows) * 3 // 4
    rows = rows[lo:hi]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    ev = []
    for s, e, c in rows:
        ev.append((s, 1, c))
        ev.append((e, -1, c))
    ev.sort()
    running = defaultdict(int)
    n = 0
    last = ev[0][0]
    by_n = defaultdict(int)          # ns with n kernels running
    by_set = defaultdict(int)        # ns per set of classes running
    cls_union = defaultdict(int)     # ns with at least one kernel of the class running
    for t, d, c in ev:
        dt = t - last
        if dt > 0:
            by_n[n] += dt
            active = tuple(sorted(k for k, v in running.items() if v > 0))
            by_set[active] += dt
            for k in active:
                cls_union[k] += dt
        running[c] += d
        n += d
        last = t
    wall = t1 - t0
    out = {
        'kernels': len(rows), 'wall_ms': wall / 1e6,
        'sum_of_durations_ms': sum(e - s for s, e, _ in rows) / 1e6,
        'time_with_n_kernels_running': {str(k): round(v / wall, 4) for k, v in sorted(by_n.items())},
        'mean_kernels_running': sum(k * v for k, v in by_n.items()) / wall,
        'time_with_class_running': {k: round(v / wall, 4) for k, v in sorted(cls_union.items(), key=lambda kv: -kv[1])},
        'top_class_sets': [{'classes': list(k), 'share': round(v / wall, 4)} for k, v in sorted(by_set.items(), key=lambda kv: -kv[1])[:12]],
    }
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
