"""DEV: scheduling study of the arc pass for tiles with MORE items than lanes (coils at 20 and at 100 slices), on the dump of the
emulation built with -DLR2_EMU_DUMP (see arc_sched_sim.py for the build line):
   SASA_EMU_SO=/tmp/libsasa_emu_dump.so LR2_EMU_DUMP=/tmp/dump.txt EMU_LR2_PRUNE=0 python -c "... run_batch(True, xyz, r, resolution=100)"
   python tools/dev/arc_sched_sim_queue.py /tmp/dump.txt
Policies: the shipped queue (refill threshold x arc steps per look), rounds of 64 items in queue order, items split into
chunks of at most g arcs (merged afterwards).  Costs in wave instructions: arc step 50, item switch 105, a look 6, a merge 40
(DESIGN.md 5).  Round 6: every policy within 8 % of the shipped one - profiles/r06_lr100_arc_schedule_sim.txt."""
import sys, math
C_ARC, C_REFILL, C_ITER, C_MERGE = 50, 105, 6, 40
def tiles(path):
    cur = None
    for line in open(path):
        f = line.split()
        if f[0] == "T":
            if cur: yield cur
            cur = dict(nq=int(f[1]), na=int(f[2]), mwt=int(f[3]), cover=int(f[4]), items=[])
        else:
            mask = 0
            for k, w in enumerate(f[2:]): mask |= int(w, 16) << (32 * k)
            cur["items"].append(bin(mask).count("1"))
    if cur: yield cur

def queue(items, refill=24, steps=2):
    nq = len(items)
    lanes = [items[i] if i < nq else 0 for i in range(64)]
    nxt = 64; cost = 0; its = 0; refills = 0
    while True:
        due = refill if nxt < nq else 64
        while True:
            waiting = sum(1 for w in lanes if w == 0)
            cost += C_ITER
            if waiting >= due: break
            for s in range(steps):
                if any(w > 0 for w in lanes): cost += C_ARC; its += 1
                lanes = [w - 1 if w > 0 else 0 for w in lanes]
        refills += 1; cost += C_REFILL
        for l in range(64):
            if lanes[l] == 0 and nxt < nq: lanes[l] = items[nxt]; nxt += 1
        if nxt >= nq and all(w == 0 for w in lanes): break
    return cost, its, refills

def rounds_sorted(items):
    """rounds of 64 items, sorted descending (the queue's order), each round runs max arcs; one switch per round"""
    cost = its = 0
    for r in range(0, len(items), 64):
        m = max(items[r:r+64]); its += m; cost += m * C_ARC + C_REFILL
    return cost, its, (len(items) + 63) // 64

def rounds_split(items, g):
    """chunks of <= g arcs (an item of c arcs -> ceil(c/g) chunks of near equal size), sorted descending, rounds of 64 chunks,
       merges: per round one merge step costed when any chunk in the round belongs to a split item"""
    ch = []
    for c in items:
        n = math.ceil(c / g)
        for k in range(n): ch.append((c * (k + 1)) // n - (c * k) // n)
    ch.sort(reverse=True)
    cost = its = 0
    for r in range(0, len(ch), 64):
        m = ch[r]; its += m; cost += m * C_ARC + C_REFILL
    nsplit = sum(1 for c in items if c > g)
    cost += C_MERGE * math.ceil(math.log2(max(2, max(math.ceil(c / g) for c in items)))) if nsplit else 0
    return cost, its, (len(ch) + 63) // 64

ts = [t for t in tiles(sys.argv[1]) if t["nq"] > 0]
n = len(ts)
arcs = sum(sum(t["items"]) for t in ts) / n
print(f"tiles {n} items/tile {sum(t['nq'] for t in ts)/n:.1f} arcs/tile {arcs:.0f} ideal trips {arcs/64:.1f}  max item {max(max(t['items']) for t in ts)}")
for rf in (8, 16, 24, 32, 48):
    for st in (1, 2, 4):
        r = [queue(t["items"], rf, st) for t in ts]
        print(f"queue refill {rf:2d} steps {st}: cost {sum(x[0] for x in r)/n:7.0f} trips {sum(x[1] for x in r)/n:5.1f} refills {sum(x[2] for x in r)/n:4.1f}")
r = [rounds_sorted(t["items"]) for t in ts]
print(f"rounds sorted: cost {sum(x[0] for x in r)/n:7.0f} trips {sum(x[1] for x in r)/n:5.1f} rounds {sum(x[2] for x in r)/n:4.1f}")
for g in (4, 6, 8, 10, 12, 16):
    r = [rounds_split(t["items"], g) for t in ts]
    print(f"rounds split g={g:2d}: cost {sum(x[0] for x in r)/n:7.0f} trips {sum(x[1] for x in r)/n:5.1f} rounds {sum(x[2] for x in r)/n:4.1f}")
