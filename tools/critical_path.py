"""Where could the step time go?  From the per-record isolated timings of a launch plan (bench.py --dump-plan-timing) and
the plan's stream semantics (slot per record, FORK / JOIN / DEP records) compute
  - the summed kernel time (what one stream would need),
  - the critical path: the longest chain of dependent records when every stream slot runs concurrently without contention,
  - per-slot busy time and the records on the critical path grouped by kind.
The measured step lies between the critical path (perfect overlap, no contention) and the sum (no overlap).

    python tools/critical_path.py gpurun_out/plan_timing.json
"""
import json
import sys
from collections import defaultdict

FORK, JOIN, DEP = 16, 17, 18


def analyse(rows):
    ready = defaultdict(float)          # time at which each slot is free / its last record is done
    owner = {}                          # slot -> index of the record that set `ready` (for back-tracking)
    pred = {}                           # record index -> predecessor record index on the critical chain
    busy = defaultdict(float)
    for k, r in enumerate(rows):
        kind, slot = r['kind'], r['slot']
        if kind == FORK:                # side slots named in the mask wait for slot 0
            for s in range(1, 4):
                if r['i0'] & (1 << (s - 1)) and ready[0] > ready[s]:
                    ready[s], owner[s] = ready[0], owner.get(0)
            continue
        if kind == JOIN:                # slot 0 waits for the side slots
            for s in range(1, 4):
                if r['i0'] & (1 << (s - 1)) and ready[s] > ready[0]:
                    ready[0], owner[0] = ready[s], owner.get(s)
            continue
        if kind == DEP:                 # i0 = source slot, i1 = destination slot
            src, dst = r['i0'], r['i1']
            if ready[src] > ready[dst]:
                ready[dst], owner[dst] = ready[src], owner.get(src)
            continue
        pred[k] = owner.get(slot)
        ready[slot] += r['ms']
        busy[slot] += r['ms']
        owner[slot] = k
    end_slot = max(ready, key=lambda s: ready[s])
    chain, k = [], owner.get(end_slot)
    while k is not None:
        chain.append(k)
        k = pred.get(k)
    return ready[end_slot], busy, chain[::-1]


def main():
    data = json.load(open(sys.argv[1]))
    total_cp = total_sum = 0.0
    for name in ('forward', 'backward'):
        rows = data[name]
        cp, busy, chain = analyse(rows)
        s = sum(r['ms'] for r in rows if r['kind'] not in (FORK, JOIN, DEP))
        total_cp += cp
        total_sum += s
        print('%-8s records %5d   sum %7.2f ms   critical path %7.2f ms   (%.0f %% of the sum)' % (name, len(rows), s, cp, 100 * cp / s))
        print('         busy per slot: ' + '  '.join('%d:%.2f' % (sl, busy[sl]) for sl in sorted(busy)))
        by = defaultdict(float)
        for k in chain:
            by[rows[k]['label'].split(' ')[0]] += rows[k]['ms']
        print('         on the critical path: ' + ', '.join('%s %.2f' % kv for kv in sorted(by.items(), key=lambda kv: -kv[1])[:8]))
    print('plans together: sum %.2f ms, critical path %.2f ms (head, losses and optimizer are outside the plans)' % (total_sum, total_cp))


if __name__ == '__main__':
    main()
