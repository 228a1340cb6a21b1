#!/usr/bin/env python3
"""Summarise the LAST `--steps` training steps of a rocprofv3 --kernel-trace CSV: per kernel name the
launches per step and the average duration, i.e. the steady-state profile without the burn-in.
Usage: python tools/trace_tail.py <kernel_trace.csv> --anchor composite_train_forward --steps 100 [--out summary.csv]"""
import argparse
import collections
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--anchor", default="composite_train_forward", help="kernel launched exactly once per step")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--out", default="")
    ap.add_argument("--timeline", default="", help="write a per-launch timeline (start, duration, queue, gap) of the "
                                                   "last --timeline-steps steps to this file")
    ap.add_argument("--timeline-steps", type=int, default=18)
    ap.add_argument("--split", default="", help="kernel-name fragment that marks a special step (e.g. grid_update_kernel = "
                                                "the density-grid refresh): regular and special steps get a table each")
    args = ap.parse_args()
    rows = []
    with open(args.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         r.get("Queue_Id", r.get("Stream_Id", "?"))))
    rows.sort()
    anchors = [i for i, r in enumerate(rows) if args.anchor in r[2]]
    if len(anchors) <= args.steps:
        raise SystemExit("not enough steps in the trace")
    first = anchors[-args.steps - 1]
    last = anchors[-1]
    sel = rows[first:last]
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n, _ in sel:
        agg[n][0] += 1
        agg[n][1] += e - s
    wall = (rows[last][0] - rows[first][0]) / args.steps / 1e3
    busy = sum(v[1] for v in agg.values()) / args.steps / 1e3
    out = [("kernel", "launches_per_step", "avg_us", "us_per_step")]
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append((n[:110], round(c / args.steps, 2), round(t / c / 1e3, 2), round(t / args.steps / 1e3, 2)))
    print(f"steps {args.steps}: wall {wall:.1f} us/step, kernel-busy {busy:.1f} us/step, "
          f"{sum(v[0] for v in agg.values()) / args.steps:.0f} launches/step")
    for row in out[:40]:
        print(*row, sep=" | ")
    if args.timeline:
        timeline(rows, anchors, args.timeline_steps, args.timeline)
    extra = []
    if args.split:
        # one window per step: from a step's anchor launch to the next one's.  The anchor (the compositor step) sits in the
        # middle of a step, so a window holds the second half of step i and the first half of step i + 1 -- a window is
        # "special" when the marker kernel runs inside it (the refresh runs in front of its step's forward).
        kinds = {"regular": collections.defaultdict(lambda: [0, 0]), "special": collections.defaultdict(lambda: [0, 0])}
        count, wall_k = {"regular": 0, "special": 0}, {"regular": 0, "special": 0}
        for a, b in zip(anchors[-args.steps - 1:-1], anchors[-args.steps:]):
            win = rows[a:b]
            kind = "special" if any(args.split in r[2] for r in win) else "regular"
            count[kind] += 1
            wall_k[kind] += rows[b][0] - rows[a][0]
            for s_, e_, n_, _ in win:
                kinds[kind][n_][0] += 1
                kinds[kind][n_][1] += e_ - s_
        for kind in ("regular", "special"):
            if not count[kind]:
                continue
            head = (f"# {kind} steps ({count[kind]} of {args.steps}; special = a window with {args.split}): wall "
                    f"{wall_k[kind] / count[kind] / 1e3:.1f} us/step, kernel-busy "
                    f"{sum(v[1] for v in kinds[kind].values()) / count[kind] / 1e3:.1f} us/step")
            print(head)
            extra.append([head])
            extra.append(["kernel", "launches_per_step", "avg_us", "us_per_step"])
            for n_, (c, t) in sorted(kinds[kind].items(), key=lambda kv: -kv[1][1]):
                row = (n_[:110], round(c / count[kind], 2), round(t / c / 1e3, 2), round(t / count[kind] / 1e3, 2))
                extra.append(list(row))
                if len([r for r in extra if len(r) == 4]) % 1000 < 30:
                    print(*row, sep=" | ")
    if args.out:
        with open(args.out, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow([f"# last {args.steps} steps: wall {wall:.1f} us/step, kernel-busy {busy:.1f} us/step"])
            w.writerows(out)
            w.writerows(extra)


def timeline(rows, anchors, steps, path):
    """One line per launch: time since the window start, duration, queue, idle time since the previous launch ended
    on the same queue."""
    first = anchors[-steps - 1]
    t0 = rows[first][0]
    last_end = {}
    with open(path, "w") as f:
        f.write("start_us,dur_us,queue,gap_us,kernel\n")
        for s, e, n, q in rows[first:anchors[-1]]:
            gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
            last_end[q] = max(e, last_end.get(q, 0))
            f.write(f"{(s - t0) / 1e3:.1f},{(e - s) / 1e3:.1f},{q},{gap:.1f},{n[:60]}\n")


if __name__ == "__main__":
    main()
